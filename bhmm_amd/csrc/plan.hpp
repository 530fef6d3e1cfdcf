// plan.hpp -- host-side planning of the time decomposition: chunk tables of the N <= 8 family
// (bhmm_amd.hip) and segment tables of the 9..64-state family (wide_api.hip).  Pure C++ (no HIP), so
// the same code runs under -fsanitize=address,undefined in the CPU sanitizer build
// (oracle/Makefile `asan`); the .hip files only allocate and upload what these functions return.
#pragma once
#include <math.h>
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace bhmm {
namespace plan {

struct ChunkPlan {
    int L = 0, Lmax = 1, G = 0, Gp = 0, chunk_mult = 1;
    std::vector<int32_t> ctraj, clen, traj_c0; // [Gp], [Gp], [K+1]
    std::vector<int64_t> ct0, cgoff;           // [Gp], [Gp]
    // two-level stitch: groups of R consecutive chunks (empty when every trajectory is short)
    int nG = 0;
    std::vector<int32_t> g0, g1, gt; // [nG], [nG], [K+1]
};

// Every trajectory is cut into ceil(T/L) chunks whose lengths differ by at most one.
//   offsets[K+1]  trajectory offsets in time steps;  N  padded state count (2, 4, 8)
//   chunk         chunk length, or <= 0 for the automatic plan
//   allow_mult    automatic plan may take two / three times the default chunk count
//   block         chunks per workgroup (the tables are padded to a multiple of it)
//   half          automatic plan with half the default chunk count (see default_chunk_count)
// Returns false if the plan would exceed 2^30 chunks.
inline int64_t default_chunk_count(int N) { return 32768 * 4 / std::max(1, N / 2); } // N/2 lanes per chunk
inline bool plan_chunks(const std::vector<int64_t> &offsets, int K, int N, int64_t total, int chunk,
                        bool allow_mult, int block, ChunkPlan &p, bool half = false)
{
    p = ChunkPlan();
    int L = chunk;
    if (L <= 0) {
        // k_estep uses N/2 lanes per chunk: 32768 chunks (N = 8) put two 64-lane wavefronts on every
        // SIMD of the 256 CUs.  Fewer, longer chunks amortise the warm-up of the speculative
        // boundaries (W / L extra steps); more chunks only help occupancy (measured optimum on
        // configs[1]: profiles/r01).
        // `half`: 16384 chunks -- ONE wavefront per SIMD in the backward sweep, which issues at 0.85-0.9
        // of the rate of two.  The better plan when the default one's chunks come out shorter than
        // about 1.6 warm-ups (a batch of a few million steps): each chunk's two warm-ups then cost
        // more than the lost occupancy (tools/chunk_scan.py; the host re-plans once the warm-up of
        // the model has been measured, bhmm_amd.hip: replan_for_warmup)
        const int64_t target = default_chunk_count(N) / (half ? 2 : 1);
        int64_t l = (total + target - 1) / target;
        // Very long chunks: two or three times as many.  The sweeps without xi accumulators (P1,
        // the forward-only pass) then have four to six long wavefronts per SIMD instead of two
        // (configs[2], 1024 x 1e6: P1 7.2 -> 5.9 ms, E-step 20.2 -> 18.7 ms), while the warm-up
        // stays below a few per cent of the chunk even if it calibrates to four times the default.
        if (l >= 3 * 9216 && allow_mult) {
            l = (l + 2) / 3;
            p.chunk_mult = 3;
        } else if (l >= 2 * 9216 && allow_mult) {
            l = (l + 1) / 2;
            p.chunk_mult = 2;
        }
        L = (int)std::min<int64_t>(std::max<int64_t>(l, 32), (int64_t)1 << 20);
    }
    p.L = L;
    p.traj_c0.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        const int64_t T = offsets[k + 1] - offsets[k];
        p.traj_c0[k] = (int32_t)p.ctraj.size();
        if (T <= 0)
            continue;
        const int64_t nck = (T + L - 1) / L;
        const int64_t base = T / nck, rem = T % nck;
        if ((int64_t)p.ctraj.size() + nck > (int64_t)1 << 30)
            return false;
        for (int64_t q = 0; q < nck; ++q) {
            const int64_t len = base + (q < rem ? 1 : 0);
            const int64_t t0 = q * base + std::min(q, rem);
            p.ctraj.push_back(k);
            p.clen.push_back((int32_t)len);
            p.ct0.push_back(t0);
            p.cgoff.push_back(offsets[k] + t0);
            p.Lmax = std::max<int>(p.Lmax, (int)len);
        }
    }
    p.traj_c0[K] = (int32_t)p.ctraj.size();
    p.G = (int)p.ctraj.size();
    p.Gp = std::max(block, (p.G + block - 1) / block * block);
    p.ctraj.resize(p.Gp, 0);
    p.clen.resize(p.Gp, 0);
    p.ct0.resize(p.Gp, 1);
    p.cgoff.resize(p.Gp, 0);
    // two-level stitch: groups of R consecutive chunks; serial depth 2R + n/R instead of n
    int nmax = 0;
    for (int k = 0; k < K; ++k)
        nmax = std::max(nmax, p.traj_c0[k + 1] - p.traj_c0[k]);
    if (nmax > 48) {
        const int R = std::max(4, std::min(256, (int)lround(sqrt(0.5 * nmax))));
        p.gt.assign(K + 1, 0);
        for (int k = 0; k < K; ++k) {
            p.gt[k] = (int32_t)p.g0.size();
            for (int cc = p.traj_c0[k]; cc < p.traj_c0[k + 1]; cc += R) {
                p.g0.push_back(cc);
                p.g1.push_back(std::min(cc + R, p.traj_c0[k + 1]));
            }
        }
        p.gt[K] = (int32_t)p.g0.size();
        p.nG = (int)p.g0.size();
    }
    return true;
}

struct SegPlan {
    std::vector<int32_t> traj, len, traj0; // [ns], [ns], [K+1]
    std::vector<int64_t> t0;               // [ns]
};

// Segments of at most seglen steps (seglen <= 0: one per trajectory), `mult` times as many;
// boundaries at multiples of four (the lazily scaled kernels rescale on t % 4 == 3).
inline void plan_segments(const std::vector<int64_t> &offsets, int K, int64_t seglen, int mult,
                          SegPlan &s)
{
    s = SegPlan();
    s.traj0.assign(K + 1, 0);
    for (int k = 0; k < K; ++k) {
        s.traj0[k] = (int32_t)s.traj.size();
        const int64_t T = offsets[k + 1] - offsets[k];
        if (T <= 0)
            continue;
        const int64_t ns = (seglen > 0 ? (T + seglen - 1) / seglen : 1) * mult;
        int64_t prev = 0;
        for (int64_t q = 1; q <= ns; ++q) {
            const int64_t b = q == ns ? T : ((q * T) / ns) & ~(int64_t)3;
            if (b <= prev)
                continue;
            s.traj.push_back(k);
            s.len.push_back((int32_t)(b - prev));
            s.t0.push_back(prev);
            prev = b;
        }
    }
    s.traj0[K] = (int32_t)s.traj.size();
}

// Tiles of the row-batched kernels (tile_kernels.hpp): 16 segments per workgroup, which runs as long
// as its longest row and takes its fast paths where all 16 rows are in the same phase -- so segments
// without a warm-up in the direction of the pass (backward = false: those that start a trajectory;
// backward = true: those that end one) get tiles of their own, and inside each class the segments are
// sorted by length (stable, longest first).  Empty slots: -1.
inline void plan_tiles(const SegPlan &s, const std::vector<int64_t> &offsets, bool backward,
                       std::vector<int32_t> &tile_seg)
{
    tile_seg.clear();
    for (int cls = 0; cls < 2; ++cls) {
        std::vector<int32_t> order;
        for (size_t i = 0; i < s.len.size(); ++i) {
            if (s.len[i] <= 0)
                continue;
            const int64_t T = offsets[s.traj[i] + 1] - offsets[s.traj[i]];
            const bool edge = backward ? s.t0[i] + s.len[i] >= T : s.t0[i] == 0;
            if ((edge ? 0 : 1) == cls)
                order.push_back((int32_t)i);
        }
        std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return s.len[a] > s.len[b]; });
        const size_t base = tile_seg.size();
        tile_seg.resize(base + (order.size() + 15) / 16 * 16, -1);
        std::copy(order.begin(), order.end(), tile_seg.begin() + base);
    }
}

// For every segment of the plan with `seglen`: the start of a segment of the twice-as-fine plan
// strictly inside it (-1: none).  Cuts the trajectories exactly like plan_segments(.., 1).
inline void plan_forward_mids(const std::vector<int64_t> &offsets, int K, int64_t seglen,
                              std::vector<int64_t> &mid)
{
    mid.clear();
    for (int k = 0; k < K; ++k) {
        const int64_t T = offsets[k + 1] - offsets[k];
        if (T <= 0)
            continue;
        const int64_t ns = (T + seglen - 1) / seglen;
        int64_t prev = 0;
        for (int64_t q = 1; q <= ns; ++q) {
            const int64_t b = q == ns ? T : ((q * T) / ns) & ~(int64_t)3;
            if (b <= prev)
                continue;
            const int64_t m2 = (((2 * q - 1) * T) / (2 * ns)) & ~(int64_t)3;
            mid.push_back(m2 > prev && m2 < b ? m2 : -1);
            prev = b;
        }
    }
}

} // namespace plan
} // namespace bhmm
