// acos over a block of doubles, four lanes at a time (AVX2, no FMA: the same operation sequence as the scalar form below).
// The trackers' velocity-direction terms (OC-SORT's OCM: 130 x 140 pairs per frame) spend a third of
// their time in libm's acos; this is the classic rational approximation (Sun's fdlibm e_acos.c: |error| < 1 ulp) written so
// that all three argument ranges share one polynomial evaluation and are blended by lane masks.
//   |x| < 0.5 : acos(x) = pi/2 - (x + x R(x^2))
//   x <= -0.5 : acos(x) = pi - 2 (s + s R(z)),            z = (1 + x) / 2, s = sqrt(z)
//   x >=  0.5 : acos(x) = 2 (df + (s R(z) + (z - df^2) / (s + df))), z = (1 - x) / 2, s = sqrt(z), df = s with its low word cleared
// with R(z) = z P(z) / Q(z). Arguments are expected in [-1, 1] (the callers clamp); +-1 give 0 and pi exactly.
// Checked against std::acos over the whole range in csrc/hosttest/sanitize_host.cpp (<= 2 ulp of pi).
#pragma once
#include <immintrin.h>

#include <cstdint>

namespace gtx {

inline void acos_block(const double* x, double* out, int n4) {   // n4 % 4 == 0 (callers pad their rows)
  const __m256d half = _mm256_set1_pd(0.5), one = _mm256_set1_pd(1.0), two = _mm256_set1_pd(2.0);
  const __m256d pio2_hi = _mm256_set1_pd(1.57079632679489655800e+00), pio2_lo = _mm256_set1_pd(6.12323399573676603587e-17);
  const __m256d pi = _mm256_set1_pd(3.14159265358979311600e+00);
  const __m256d pS0 = _mm256_set1_pd(1.66666666666666657415e-01), pS1 = _mm256_set1_pd(-3.25565818622400915405e-01),
                pS2 = _mm256_set1_pd(2.01212532134862925881e-01), pS3 = _mm256_set1_pd(-4.00555345006794114027e-02),
                pS4 = _mm256_set1_pd(7.91534994289814532176e-04), pS5 = _mm256_set1_pd(3.47933107596021167570e-05);
  const __m256d qS1 = _mm256_set1_pd(-2.40339491173441421878e+00), qS2 = _mm256_set1_pd(2.02094576023350569471e+00),
                qS3 = _mm256_set1_pd(-6.88283971605453293030e-01), qS4 = _mm256_set1_pd(7.70381505559019352791e-02);
  const __m256d abs_mask = _mm256_castsi256_pd(_mm256_set1_epi64x(0x7fffffffffffffffLL));
  const __m256d hi_mask = _mm256_castsi256_pd(_mm256_set1_epi64x((long long)0xffffffff00000000ULL));
  auto mad = [](__m256d a, __m256d b, __m256d c) { return _mm256_add_pd(_mm256_mul_pd(a, b), c); };
  for (int i = 0; i < n4; i += 4) {
    const __m256d v = _mm256_loadu_pd(x + i);
    const __m256d a = _mm256_and_pd(v, abs_mask);
    const __m256d small = _mm256_cmp_pd(a, half, _CMP_LT_OQ);
    const __m256d neg = _mm256_cmp_pd(v, _mm256_setzero_pd(), _CMP_LT_OQ);
    const __m256d z = _mm256_blendv_pd(_mm256_mul_pd(_mm256_sub_pd(one, a), half), _mm256_mul_pd(v, v), small);
    __m256d p = mad(z, pS5, pS4);
    p = mad(z, p, pS3);
    p = mad(z, p, pS2);
    p = mad(z, p, pS1);
    p = mad(z, p, pS0);
    p = _mm256_mul_pd(z, p);
    __m256d q = mad(z, qS4, qS3);
    q = mad(z, q, qS2);
    q = mad(z, q, qS1);
    q = mad(z, q, one);
    const __m256d r = _mm256_div_pd(p, q);
    // |x| < 0.5
    const __m256d r_small = _mm256_sub_pd(pio2_hi, _mm256_sub_pd(v, _mm256_sub_pd(pio2_lo, _mm256_mul_pd(v, r))));
    // the two outer ranges
    const __m256d s = _mm256_sqrt_pd(z);
    const __m256d w_neg = _mm256_sub_pd(_mm256_mul_pd(r, s), pio2_lo);
    const __m256d r_neg = _mm256_sub_pd(pi, _mm256_mul_pd(two, _mm256_add_pd(s, w_neg)));
    const __m256d df = _mm256_and_pd(s, hi_mask);
    const __m256d sd = _mm256_add_pd(s, df);
    const __m256d c = _mm256_div_pd(_mm256_sub_pd(z, _mm256_mul_pd(df, df)), _mm256_blendv_pd(sd, one, _mm256_cmp_pd(sd, _mm256_setzero_pd(), _CMP_EQ_OQ)));
    const __m256d w_pos = _mm256_add_pd(_mm256_mul_pd(r, s), c);
    const __m256d r_pos = _mm256_mul_pd(two, _mm256_add_pd(df, w_pos));
    const __m256d big = _mm256_blendv_pd(r_pos, r_neg, neg);
    _mm256_storeu_pd(out + i, _mm256_blendv_pd(big, r_small, small));
  }
}

}  // namespace gtx
