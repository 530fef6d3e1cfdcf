// draw_verify.hpp -- closing the last window of the batched backward draw (_hidden.c:283-305,330-378).
//
// The batched Gibbs step draws from alpha rows of a time-parallel forward pass whose chunk / segment
// boundaries were VERIFIED against their predecessors, not computed from them: such a row equals the
// serial recursion's up to the largest boundary deviation the check measured (dev <= 1e-11 by the
// check's tolerance, typically 1e-13).  A draw "first i with cumsum_i >= r" is therefore only certain
// when no cumulative sum lies within that distance of the uniform.  The sampler kernels (k_smp_maps,
// k_wide_sample_seg, k_gen_sample_seg) record every draw whose decisive gap |P_q - r S| / S is below a
// watch tolerance (64 x the deviation) as a DrawEvent; k_draw_verify then recomputes alpha_t for each
// event by the serial recursion itself (_hidden.c:16-66: per-step normalisation by the row sum) over a
// long window [t - Wlong, t] -- from TWO different start vectors, so that "the window was long enough"
// is measured (both runs must arrive at the same row to 1e-14) and not assumed; a window that reaches
// step 0 starts from pi and is the serial run outright -- and decides the draw again with the
// reference's own arithmetic (_normalize, then the first state whose cumulative sum reaches r).  If
// every event's decision stands, the path is the serial draw's; otherwise (or when a window did not
// converge) the host repeats the call on the exact alpha rows (transfer matrices / serial recursion).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bhmm {

struct DrawEvent {
    int64_t t;   // step inside the trajectory
    double r;    // the uniform of that step
    double gap;  // min_q |P_q - r S| / S as the sampler saw it (negative: not computed -- always checked)
    int32_t k;   // trajectory
    int32_t nxt; // state drawn at t + 1 (column of A); ignored at t = T - 1
    int32_t pick;
    int32_t pad_;
};

constexpr unsigned int DRAW_EVENT_CAP = 2048;

// what a sampler kernel gets: tol = 0 switches the watch off
struct DrawWatch {
    DrawEvent *ev;
    unsigned int *count; // events seen (may exceed DRAW_EVENT_CAP: overflow -> the host repeats the call)
    double tol;
};

__device__ __forceinline__ void draw_record(const DrawWatch &w, int k, int64_t t, int nxt, double r,
                                            int pick, double gap)
{
    const unsigned int q = atomicAdd(w.count, 1u);
    if (q < DRAW_EVENT_CAP) {
        DrawEvent e;
        e.t = t;
        e.r = r;
        e.gap = gap;
        e.k = k;
        e.nxt = nxt;
        e.pick = pick;
        e.pad_ = 0;
        w.ev[q] = e;
    }
}

// result[0] += events whose decision does not stand; result[1] += events whose window did not converge
// (or met a vector that cannot be normalised); result[2] += events looked at (gap <= thr: a kernel that
// cannot know the measured deviation at launch records with a static tolerance, the filter is applied here).
// One workgroup per event, any n <= 1024.
//   model: A [n][n] | pi [n] | par0 | par1   (gaussian: mu [n], sigma [n]; discrete: B [n][M] row-major)
//   obs_rm: trajectory-major observations of the context (double / int32 / n doubles per step)
// kind: 0 gaussian, 1 discrete, 2 explicit (the EMIT_* values of estep_kernels.hpp)
[[maybe_unused]] static __global__ __launch_bounds__(256) void k_draw_verify(const DrawEvent *ev, int nev, const double *model, int n,
                                                     int M, int kind, const void *obs_rm, const int64_t *off,
                                                     int64_t Wlong, double thr, unsigned int *result)
{
    extern __shared__ double dv_sm[]; // x0 | x1 | y0 | y1 : [n] each; red [8]
    double *x0 = dv_sm, *x1 = dv_sm + n, *y0 = dv_sm + 2 * n, *y1 = dv_sm + 3 * n, *red = dv_sm + 4 * n;
    const int tid = threadIdx.x;
    if ((int)blockIdx.x >= nev)
        return;
    const DrawEvent e = ev[blockIdx.x];
    if (e.gap > thr)
        return;
    if (tid == 0)
        atomicAdd(&result[2], 1u);
    const double *A = model, *pi = model + (int64_t)n * n, *par0 = pi + n;
    const double *par1 = par0 + (kind == 1 ? (int64_t)n * M : n);
    const int64_t o0 = off[e.k], T = off[e.k + 1] - o0;
    const int64_t ws = e.t > Wlong ? e.t - Wlong : 0;
    auto block_sum2 = [&](double a, double b, double &sa, double &sb) {
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, 64);
            b += __shfl_xor(b, o, 64);
        }
        __syncthreads(); // (red may still be read from the previous use)
        if ((tid & 63) == 0) {
            red[tid >> 6] = a;
            red[4 + (tid >> 6)] = b;
        }
        __syncthreads();
        sa = (red[0] + red[1]) + (red[2] + red[3]);
        sb = (red[4] + red[5]) + (red[6] + red[7]);
    };
    // start vectors of a window that does not reach step 0: uniform, and a ramp (both positive everywhere)
    for (int j = tid; j < n; j += 256) {
        x0[j] = 1.0 / (double)n;
        x1[j] = 2.0 * (double)(j + 1) / ((double)n * (double)(n + 1));
    }
    __syncthreads();
    bool bad = false;
    for (int64_t s = ws; s <= e.t; ++s) {
        // emission row of step s (_gaussian.c:5-21 / discrete.py:130-157 / the caller's pobs row)
        double p[4];
        bool nz = false;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = tid + 256 * q;
            p[q] = 0.0;
            if (j < n) {
                if (kind == 0) {
                    const double o = static_cast<const double *>(obs_rm)[o0 + s];
                    const double sg = par1[j];
                    const double C = 1.0 / (sqrt(2.0 * M_PI) * sg);
                    const double d = (o - par0[j]) / sg;
                    p[q] = C * exp(-0.5 * d * d);
                } else if (kind == 1) {
                    const int sym = static_cast<const int32_t *>(obs_rm)[o0 + s];
                    p[q] = par0[(int64_t)j * M + sym];
                } else {
                    p[q] = static_cast<const double *>(obs_rm)[(o0 + s) * n + j];
                }
                nz |= p[q] != 0.0;
            }
        }
        if (kind == 0 && !__syncthreads_or(nz ? 1 : 0)) { // outlier rule (outputmodel.py:126-130)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                p[q] = 1.0;
        }
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = tid + 256 * q;
            if (j < n) {
                double a0, a1;
                if (s == 0) {
                    a0 = a1 = pi[j] * p[q]; // _hidden.c:27-31
                } else {
                    a0 = a1 = 0.0;
                    for (int i = 0; i < n; ++i) { // _hidden.c:44-48, ascending i
                        const double aij = A[(int64_t)i * n + j];
                        a0 += x0[i] * aij;
                        a1 += x1[i] * aij;
                    }
                    a0 *= p[q];
                    a1 *= p[q];
                }
                y0[j] = a0;
                y1[j] = a1;
                sa += a0;
                sb += a1;
            }
        }
        double c0, c1;
        block_sum2(sa, sb, c0, c1);
        if (!(c0 > 0.0) || !(c1 > 0.0) || !(c0 < 1e300) || !(c1 < 1e300)) {
            bad = true; // (uniform over the workgroup)
            break;
        }
        for (int j = tid; j < n; j += 256) {
            x0[j] = y0[j] / c0;
            x1[j] = y1[j] / c1;
        }
        __syncthreads();
    }
    if (tid != 0)
        return;
    if (bad) {
        atomicAdd(&result[1], 1u);
        return;
    }
    // both windows must have arrived at the same row
    double dev = 0.0;
    for (int i = 0; i < n; ++i)
        dev = fmax(dev, fabs(x0[i] - x1[i]));
    if (!(dev <= 1e-14)) {
        atomicAdd(&result[1], 1u);
        return;
    }
    // the reference's draw on that row (_hidden.c:347-372 with _normalize and _random_choice :283-319)
    auto decide = [&](const double *x) {
        const bool last = e.t == T - 1;
        double S = 0.0;
        for (int i = 0; i < n; ++i) {
            y0[i] = last ? x[i] : x[i] * A[(int64_t)i * n + e.nxt];
            S += y0[i];
        }
        double acc = 0.0;
        int pick = -1;
        for (int i = 0; i < n; ++i) {
            acc += y0[i] / S;
            if (pick < 0 && acc >= e.r)
                pick = i;
        }
        return pick;
    };
    const int p0 = decide(x0), p1 = decide(x1);
    if (p0 != p1)
        atomicAdd(&result[1], 1u);
    else if (p0 != e.pick)
        atomicAdd(&result[0], 1u);
}

} // namespace bhmm
