// ctx.hpp -- host-side state of the batched engine (opaque to C callers as bhmm_ctx).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>
#include <string>
#include <vector>

#include "../../include/bhmm_amd.h"

namespace bhmm {

void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what);

#define BHMM_HIP(call)                                  \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess)                           \
            return ::bhmm::hip_fail(e_, #call);         \
    } while (0)

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete; // (owns its allocation)
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); } // (bhmm_ctx_destroy deletes the context on its own device: nothing is left behind)
    int ensure(size_t count)
    {
        if (count <= n)
            return BHMM_OK;
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
        hipError_t e = hipMalloc(reinterpret_cast<void **>(&p), count * sizeof(T));
        if (e != hipSuccess) {
            set_error(std::string("hipMalloc of ") + std::to_string(count * sizeof(T)) +
                      " bytes failed: " + hipGetErrorString(e));
            (void)hipGetLastError();
            return BHMM_ERR_NO_MEM;
        }
        n = count;
        // debugging aid: BHMM_AMD_POISON=1 fills every fresh allocation with 0xFF bytes (NaN doubles,
        // -1 integers), so that a kernel that reads what nothing wrote shows up at once instead of
        // depending on what the allocator handed out
        static const bool poison = getenv("BHMM_AMD_POISON") != nullptr;
        if (poison) { // (the fill runs on the null stream; the context's streams do not wait for it by themselves)
            (void)hipMemset(p, 0xFF, count * sizeof(T));
            (void)hipDeviceSynchronize();
        }
        return BHMM_OK;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};

} // namespace bhmm

struct bhmm_ctx {
    int device = 0;
    int num_simd = 1024; // 4 per compute unit (set at creation)
    hipStream_t stream = nullptr;
    bool own_stream = false;

    // loaded problem
    int kind = -1;
    int n = 0;    // real number of states
    int N = 0;    // padded (2, 4, 8)
    int M = 0;    // symbols
    bool bt_global = false; // discrete alphabet too large for the LDS tables (global / L2 instead)
    int K = 0;    // trajectories
    int64_t total = 0;
    int L = 0, Lmax = 0, G = 0, Gp = 0;
    int chunk_mult = 1; // automatic plan: 2 / 3 times the default chunk count (very long chunks)
    std::vector<int64_t> offsets;  // [K+1]
    std::vector<int32_t> traj_c0;  // [K+1] first chunk of each trajectory

    // device buffers
    bhmm::DevBuf<int32_t> d_ctraj, d_clen, d_traj_c0;
    bhmm::DevBuf<int64_t> d_ct0, d_cgoff;
    bhmm::DevBuf<char> d_obs_ci;     // CI observations (double / int32 / N doubles)
    bhmm::DevBuf<char> d_obs_rm;     // trajectory-major observations (Viterbi / sampling)
    bhmm::DevBuf<double> d_Bt;       // [M][N] transposed emission matrix
    bhmm::DevBuf<double> d_M;        // chunk transfer matrices
    bhmm::DevBuf<double> d_aentry, d_bexit, d_ws, d_gamma_ci;
    bhmm::DevBuf<float> d_ws32;      // Gibbs step: CI alpha rows rounded to fp32 (forward-only pass)
    bool rows32_valid = false;       // ... written by the last forward-only pass
    bool fwd_defer = false;          // forward-only pass: enqueue the boundary check, do not wait for it
    bool fwd_pending = false;        // ... its verdict is still to be read (forward_ci_verdict)
    bhmm::DevBuf<double> d_logLc, d_logLk, d_gamma0, d_partials, d_dpartials, d_stats;
    bhmm::DevBuf<char> d_scratch;    // paths, uniforms, pointer tables ...
    bhmm::DevBuf<char> d_scratch2;
    bhmm::DevBuf<int64_t> d_offsets; // [K+1] trajectory offsets (time steps)
    bhmm::DevBuf<int32_t> d_obsnan;  // [1] set by the upload if a gaussian observation is NaN
    bhmm::DevBuf<int64_t> d_soff;    // [K] position of each trajectory in the device random stream
                                     // (unset: its offset here; a sharded caller sets global ones)
    bhmm::DevBuf<double> d_Brm;      // [n][M] emission matrix, row-major (path kernels)
    bhmm::DevBuf<double> d_alpha_rm; // [total][n] alpha, trajectory-major (path sampling)
    bhmm::DevBuf<double> d_wmodel;   // model parameters of the 9..64-state family
    // speculative (verified) chunk boundaries, see k_estep<..., SPEC> (estep_sweep.hpp) / k_spec_check
    bool spec_enabled = true;
    int spec_W = 288;             // warm-up length: read off the measured forgetting curve at the
                                  // first E-step on new data (probe_warmup), lengthened after a
                                  // failed verification
    bool spec_W_fixed = false;    // given by the caller (option / BHMM_AMD_SPEC_W): never probed
    bool spec_calibrated = false; // probe done for this set of observations
    int spec_probes_left = 0;     // re-probes allowed after failed checks
    bhmm::DevBuf<char> d_probe;   // probe: sample positions + forgetting curve
    int spec_fail = 0, spec_ok = 0;
    float spec_last_dev = 0.f;
    bhmm::DevBuf<double> d_aexit, d_bentry;
    // boundary vectors carried from one E-step to the next (estep_sweep.hpp: Carry)
    bool carry_enabled = true;    // option "carry" / BHMM_AMD_CARRY=0
    bool carry_valid = false;     // d_carry_* hold vectors of the previous (verified) E-step
    int carry_Wc = 0;             // ... captured for warm-ups of about this many steps
    int carry_use = 0;            // this launch: warm-ups start from the carried vectors (their Wc)
    int carry_cap = 0;            // this launch: capture beta when this many steps remain (0: none)
    int carry_Wout = 0;           // this launch: capture alpha this many steps before the chunks
    bool carry_store = false;     // this launch: store the captures (else the sweep is only split there)
    double carry_kappa = 100.0;   // boundary deviation per unit of model change, running bound
    double carry_delta = -1.0;    // model change against the previous E-step (-1: unknown)
    double carry_rdec = 0.0;      // decades of forgetting per warm-up step, measured: the deviation the
                                  // check found after a FULL warm-up from the uniform vector (0: unknown)
    int carry_ok = 0, carry_fail = 0, carry_last_W = 0;
    std::vector<double> prev_model; // [A | par0 | par1] of the previous E-step
    std::vector<double> last_pi;    // ... and its initial distribution, flags (nonfinite_retry repeats that call)
    int last_flags = 0;
    bhmm::DevBuf<double> d_carry_a, d_carry_b;
    bhmm::DevBuf<int32_t> d_carry_da, d_carry_db;
    bhmm::DevBuf<unsigned int> d_specres;
    bhmm::DevBuf<int32_t> d_ea;       // exponents of the stored alpha rows (k_estep PH_P1 -> PH_P2)
    bhmm::DevBuf<double> d_tail;      // same layout as h_raw, written by k_tail (one D2H copy)
    bhmm::DevBuf<double> d_fold;      // partial statistics folded 128 rows at a time (k_fold_rows)
    bhmm::DevBuf<double> d_tbpart;    // k_tail: per trajectory block [sum logL | sum gamma_0 (N)]
    int tail_slot = 0;                // verdict word set of the next E-step
    // 9..64 states: time segments of the Viterbi pass [0] and of the backward sampler [1] (path_api.hip)
    struct PathPlan {
        int nseg = 0;
        int64_t seglen = 0;
        int64_t maxlen = 0; // longest segment of the plan (boundaries are rounded to multiples of four: up to seglen + 3)
        bhmm::DevBuf<int32_t> traj, len, traj0; // traj0[k]: first segment of trajectory k, [K + 1]
        bhmm::DevBuf<int64_t> t0;
    } pplan[3]; // [0] Viterbi pass, [1] backward sampler, [2] back-trace of the Viterbi pass (finer: the walk is a
                // chain of dependent look-ups, its time is the length of a segment)
    bhmm::DevBuf<uint8_t> d_vmaps, d_vend; // back-trace over segments: maps [nseg][64], last state of each segment
    int smp_W = 0;                    // sampler: warm-up (steps above a segment) of the next call
    int smp_seg_mismatch = 0, smp_seg_rounds = 0;
    bool draw_fwd_segmented = false;  // ... its alpha rows came from the time-segmented forward pass
    bool smp_segmented = false;       // the last sample_paths call ran over time segments
    bhmm::DevBuf<int32_t> d_sentry, d_sexit;
    // draws decided within reach of the alpha rows' verified deviation (draw_verify.hpp)
    bhmm::DevBuf<char> d_dv;          // [count | disagree | unconverged | pad] (16 B) | DrawEvent[DRAW_EVENT_CAP] | model
    bool draw_watch = true;           // option "draw_watch": record and verify such draws
    double draw_watch_tol = 0.0;      // option "draw_watch_tol" (tests): watch tolerance instead of 64 x deviation
    bool draw_test_redo = false;      // option "draw_test_redo" (tests): treat every event as a decision that did not stand
    double draw_alpha_dev = 0.0;      // largest boundary deviation of the forward pass the draws read (0: exact rows)
    bool draw_force_exact = false;    // the call in flight is the repeat on exact alpha rows
    unsigned int draw_events = 0;     // last call: draws inside the watch tolerance
    unsigned int draw_checked = 0;    // ... of them decided again on the windowed serial recursion
    unsigned int draw_redone = 0;     // ... calls repeated on the exact alpha rows (0 / 1)
    double spec_tol = 1e-11;          // N <= 8: tolerance of the boundary check (option spec_tol)
    int vit_seg_per_simd = 2;
    int vit_seg_warmups = 2;          // a Viterbi segment is at least this many warm-ups long (measured: 1, 2, 4)
    int smp_seg_per_simd = 4;         // (the draw is a short dependent chain: more wavefronts per SIMD hide it)
    int vit_seg_mismatch = 0, vit_seg_rounds = 0;
    bool vit_margin = true;           // accept a first pass whose boundaries are equal to 1e-12 when every decision ON its
                                      // path has a margin (k_vit_margin) instead of running fix-up rounds
    bool vit_margin_want = false;     // 9..64 states: a call on these observations needed two or more fix-up rounds (the
                                      // margin acceptance costs about one short round: it is tried from then on)
    int vit_far = 0;                  // ... boundaries of the last first pass that were not equal to 1e-12
    bool vit_mend = true;             // option "viterbi_mend": those segments alone are run again up to a kept vector
    int vit_mended = 0;               // ... how many the last call ran again that way
    int vit_margin_used = 0;          // ... the last call was accepted that way
    int vit_margin_close = 0;         // ... segments with a close decision on the path in the last call (then: rounds)
    bhmm::DevBuf<double> d_vckpt;  // the first pass's vector at every 64th step
    bhmm::DevBuf<uint8_t> d_vflag; // segments the next fix-up round repeats
    bool vit_seg_given_up = false;    // ... boundaries did not coalesce on these observations: serial kernel
    int vit_rows_fail = 0;            // 129..256 states: first passes in a row that were not accepted (two: given up)
    int vit_W = 0;                    // warm-up the chunked Viterbi last verified with (0: spec_W)
    int vit_bad = 0;                  // ... a shorter one that did not verify
    bool vit_explore = true;          // ... still trying shorter ones (path_api.hip)
    bool chunk_auto = true;           // the caller left the chunk length to the library
    bool replanned_half = false;      // ... and it has been re-planned with half the chunks (once)
    bool serial_retry_done = false;   // non-finite counts: re-planned with one chunk per trajectory (once)
    unsigned int viterbi_close = 0;   // ... number of lanes that met a close decision
    bool viterbi_chunked = false;     // last bhmm_viterbi_batch ran chunk-parallel (verified)
    int wide_replans = 0;             // 9..64 states: segment plans re-made after failed checks
    bool wseg_given_up = false;       // ... and segmentation abandoned for this data set
    bool tail_ready = false;          // d_tail allocated and its verdict words cleared
    bool ev_lean = false;             // last E-step recorded only ev[2..4]
    unsigned int *h_specres = nullptr; // pinned
    double *h_small = nullptr;         // pinned, 64 KB: small results of the path calls (one copy per call)
    // two-level stitch: groups of consecutive chunks (empty when every trajectory is short)
    int nG = 0;
    bhmm::DevBuf<int32_t> d_grp_c0, d_grp_c1, d_grp_traj0; // [nG], [nG], [K+1]
    bhmm::DevBuf<double> d_P, d_agrp, d_bgrp;              // group products / boundary vectors
    bool wide = false;               // nstates > 8: wide_kernels.hpp family
    bool gen = false;                // nstates > 64: gen_kernels.hpp family (any N, trajectory-major,
                                     // one workgroup per trajectory; `wide` is false then)
    bhmm::DevBuf<double> d_gpobs, d_gW, d_gAt, d_gxipart, d_gpart, d_gsym;
    bhmm::DevBuf<double> d_bigBf, d_bigBb; // more than 128 states: A / A^T in matrix-operand order (big_kernels.hpp)
    // wide family: segment tables.  [0] = one segment per trajectory (exact serial recursion),
    // [1] = time-segmented plan with verified warm-up boundaries (optional)
    // plans: 0 = one segment per trajectory, 1 = time segments (both passes), 2 = the forward pass's
    // own, finer time segments (64 states: it fits two wavefronts per SIMD, the backward pass one)
    int w_nseg[3] = {0, 0, 0};
    bhmm::DevBuf<int32_t> d_wseg_traj[3], d_wseg_len[3], d_wseg_traj0[3];
    bhmm::DevBuf<int64_t> d_wseg_t0[3];
    bhmm::DevBuf<int64_t> d_wseg_fmid;  // plan 1: start of the forward pass's second segment inside each
    bhmm::DevBuf<double> d_wlogLseg, d_waentry, d_waexit, d_wbexit, d_wbentry;
    // row-batched matrix-core recursions (tile_kernels.hpp): 16 segments per workgroup
    bool tile_enabled = true;        // option "tile" / BHMM_AMD_TILE=0: takes effect at the next set_observations ...
    bool tile_latched = true;        // ... where it is latched: plans, buffers and launches of one data set all use THIS
    int tile_per_cu = 1;             // tiles the segment plan aims at per compute unit (option "tile_per_cu")
    int w_ntiles[3] = {0, 0, 0}, w_ntilesb[3] = {0, 0, 0};
    bhmm::DevBuf<int32_t> d_tile_seg[3];  // [16 * ntiles] segment of every tile row (-1: none), forward pass
    bhmm::DevBuf<int32_t> d_tile_segb[3]; // ... backward pass (tiles are formed per direction, plan.hpp)
    bhmm::DevBuf<int32_t> d_wexp;    // [total] exponent the forward pass removed at every step
    bhmm::DevBuf<int32_t> d_wePseg;  // [segments] ... summed over the main part of every segment
    unsigned int wide_trouble = 0;   // flag word of the last lazily scaled E-step (which self-check fired)
    int wide_retry = 0;              // time-segmented attempts of the E-step call in flight that did not verify
    int tile_settle = 0;             // warm-up refinements done for these observations (at most 4, first E-step)
    int tile_W_good = 0;             // ... the last warm-up that verified
    int tile_reason = 0;             // why the tile path was left (0: it was not): 1 calibration saw a self-check fire,
                                     // 2 calibration did not converge, 3 warm-up >= half a trajectory, 4 self-check in an
                                     // E-step, 5 boundaries did not verify after three attempts, 6 fixed warm-up does not verify
    bool tile_used = false;          // the last E-step ran on the tile kernels
    bool wseg_enabled = true;
    bool wseg_split = true;     // 64 states: own, finer plan for the forward pass (wide_plan_segments)
    int64_t wseg_cur_len = 0;   // segment length of plan 1 (re-plans only lengthen: buffers are sized once)
    int wseg_len = 0;                // 0 = automatic
    double *h_raw = nullptr;         // pinned: [verdict words, 2 sets (4 doubles) | stats | logL_k]
    double *h_pinned = nullptr;      // = h_raw + 4: stats + logL_k landing zone
    size_t h_pinned_n = 0;

    bool gamma_valid = false;
    bool gamma_wanted = false;  // the E-step in flight stores gamma (a re-plan must re-size the rows)
    bool careful = false;       // E-steps use the kernel with the per-step outlier branch
    bool careful_retry = false; // the last verdict asked for a repeat with that kernel
    bool wide_careful = false;  // 9..64 states: lazily scaled kernels left their range on these data
    bool prefetched = false;          // stats + logL_k of the last E-step already sit in h_pinned
    bool logLk_prefetched = true;     // ... logL_k included (not for many trajectories: on demand)
    bool last_stats_internal = true;
    bool last_stats_checked = false;  // the caller's buffer of the last E-step has been looked at by bhmm_estep_fetch
    double *last_stats = nullptr;     // device buffer the last E-step wrote its statistics to
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_pending = false;
    double last_ms[5] = {0, 0, 0, 0, 0};
};
