// host_rev_avx2.cpp -- the four-lane (AVX2 + FMA) instantiation of the reversible transition-matrix sampler
// (host_rev_sampler.hpp).  This translation unit alone is compiled with -mavx2 -mfma; host_model.cpp calls it
// only when the CPU reports both (sample_reversible_sweeps), else the one-lane instantiation -- the same chain.
#include <immintrin.h>

#include "host_rev_sampler.hpp"

namespace bhmm {
namespace host {
namespace revs {

struct V4 {
    static const int W = 4;
    typedef __m256d D;
    typedef __m256i U;
    typedef __m256d M; // all-ones lanes
    static D set1(double x) { return _mm256_set1_pd(x); }
    static U setu(uint64_t x) { return _mm256_set1_epi64x((long long)x); }
    static D add(D a, D b) { return _mm256_add_pd(a, b); }
    static D sub(D a, D b) { return _mm256_sub_pd(a, b); }
    static D mul(D a, D b) { return _mm256_mul_pd(a, b); }
    static D div(D a, D b) { return _mm256_div_pd(a, b); }
    static D sqrt(D a) { return _mm256_sqrt_pd(a); }
    static D fma(D a, D b, D c) { return _mm256_fmadd_pd(a, b, c); }
    static D max(D a, D b) { return _mm256_max_pd(a, b); }
    static D min(D a, D b) { return _mm256_min_pd(a, b); }
    static D abs(D a) { return _mm256_andnot_pd(_mm256_set1_pd(-0.0), a); }
    static D rint(D a) { return _mm256_round_pd(a, _MM_FROUND_TO_NEAREST_INT | _MM_FROUND_NO_EXC); }
    static M lt(D a, D b) { return _mm256_cmp_pd(a, b, _CMP_LT_OQ); }
    static M gt(D a, D b) { return _mm256_cmp_pd(a, b, _CMP_GT_OQ); }
    static M ge(D a, D b) { return _mm256_cmp_pd(a, b, _CMP_GE_OQ); }
    static M eq(D a, D b) { return _mm256_cmp_pd(a, b, _CMP_EQ_OQ); }
    static M andm(M a, M b) { return _mm256_and_pd(a, b); }
    static M orm(M a, M b) { return _mm256_or_pd(a, b); }
    static M notm(M a) { return _mm256_xor_pd(a, _mm256_castsi256_pd(_mm256_set1_epi64x(-1))); }
    static bool any(M a) { return _mm256_movemask_pd(a) != 0; }
    static D blend(M m, D a, D b) { return _mm256_blendv_pd(b, a, m); }
    static U as_u(D a) { return _mm256_castpd_si256(a); }
    static D as_d(U a) { return _mm256_castsi256_pd(a); }
    static U addu(U a, U b) { return _mm256_add_epi64(a, b); }
    static U subu(U a, U b) { return _mm256_sub_epi64(a, b); }
    static U andu(U a, U b) { return _mm256_and_si256(a, b); }
    static U oru(U a, U b) { return _mm256_or_si256(a, b); }
    static U srlu(U a, int s) { return _mm256_srl_epi64(a, _mm_cvtsi32_si128(s)); }
    static U sllu(U a, int s) { return _mm256_sll_epi64(a, _mm_cvtsi32_si128(s)); }
    static U mask_u(M m) { return _mm256_castpd_si256(m); }
    // low 64 bits of a * c: alo clo + ((alo chi + ahi clo) << 32)
    static U mulu_const(U a, uint64_t c)
    {
        const U clo = _mm256_set1_epi64x((long long)(c & 0xFFFFFFFFull)), chi = _mm256_set1_epi64x((long long)(c >> 32));
        const U ahi = _mm256_srli_epi64(a, 32);
        const U ll = _mm256_mul_epu32(a, clo);
        const U cross = _mm256_add_epi64(_mm256_mul_epu32(a, chi), _mm256_mul_epu32(ahi, clo));
        return _mm256_add_epi64(ll, _mm256_slli_epi64(cross, 32));
    }
    static U mix64(U z)
    {
        z = mulu_const(_mm256_xor_si256(z, _mm256_srli_epi64(z, 30)), 0xBF58476D1CE4E5B9ull);
        z = mulu_const(_mm256_xor_si256(z, _mm256_srli_epi64(z, 27)), 0x94D049BB133111EBull);
        return _mm256_xor_si256(z, _mm256_srli_epi64(z, 31));
    }
    static D u2d_small(U a) // a < 2^52
    {
        return _mm256_sub_pd(_mm256_castsi256_pd(_mm256_or_si256(a, _mm256_set1_epi64x(0x4330000000000000ll))),
                             _mm256_set1_pd(4503599627370496.0));
    }
    static D gather(const double *t, U idx) { return _mm256_i64gather_pd(t, idx, 8); }
    static bool lane_m(M m, int l) { return ((_mm256_movemask_pd(m) >> l) & 1) != 0; }
    static double lane_d(D a, int l)
    {
        alignas(32) double t[4];
        _mm256_store_pd(t, a);
        return t[l];
    }
    static uint64_t lane_u(U a, int l)
    {
        alignas(32) uint64_t t[4];
        _mm256_store_si256(reinterpret_cast<__m256i *>(t), a);
        return t[l];
    }
    static D set_lane_d(D a, int l, double x)
    {
        alignas(32) double t[4];
        _mm256_store_pd(t, a);
        t[l] = x;
        return _mm256_load_pd(t);
    }
    static U set_lane_u(U a, int l, uint64_t x)
    {
        alignas(32) uint64_t t[4];
        _mm256_store_si256(reinterpret_cast<__m256i *>(t), a);
        t[l] = x;
        return _mm256_load_si256(reinterpret_cast<const __m256i *>(t));
    }
    static D load_d(const double *p) { return _mm256_loadu_pd(p); }
    static U load_u(const uint64_t *p) { return _mm256_loadu_si256(reinterpret_cast<const __m256i *>(p)); }
    static M load_m(const bool *p)
    {
        return _mm256_castsi256_pd(_mm256_set_epi64x(p[3] ? -1ll : 0ll, p[2] ? -1ll : 0ll, p[1] ? -1ll : 0ll, p[0] ? -1ll : 0ll));
    }
    static void store_d(double *p, D a) { _mm256_storeu_pd(p, a); }
};

} // namespace revs

void sample_reversible_sweeps_avx2(const double *C, int n, int64_t nsweeps, uint64_t base, double *X)
{
    revs::sample_reversible_sweeps_v<revs::V4>(C, n, nsweeps, base, X);
}

} // namespace host
} // namespace bhmm
