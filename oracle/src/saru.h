/* ORACLE — TEST INFRASTRUCTURE ONLY (see common.h).
 *
 * Saru PRNG restated in C from third_party/saruprng.cuh:
 *   seeding 1/2/3 seeds   :222-273
 *   advanceLCG<1>/Weyl<1> :200-201, :215-218
 *   u32()                 :340-352
 *   f(), f(low,high)      :360-386
 *   gf() Box–Muller       :115-128       (float transcendental path: NOT bit-portable)
 *   gd()                  :130-143       (note: it calls f(), logf, sqrtf, sinf, cosf — float!)
 * All integer arithmetic is mod 2^32 exactly as in the reference, so u32 streams are bit exact.
 * parity: the reference ships no known-answer vector for Saru ("parity unpinned" for the raw
 * stream); it is pinned indirectly by the statistical checks in tests/test_oracle_saru.py.
 */
#ifndef ORACLE_SARU_H
#define ORACLE_SARU_H
#include <float.h>
#include <math.h>
#include <stdint.h>

typedef struct { uint32_t state, wstate; } Saru;

#define SARU_LCGA 0x4beb5d59u
#define SARU_LCGC 0x2600e1f7u
#define SARU_oWeylPeriod 0xda879addu
#define SARU_oWeylOffset 0x8009d14bu

static inline int32_t saru_s(uint32_t x) { return (int32_t)x; }

static inline void saru_finish_seed(Saru *s, uint32_t seed1, uint32_t seed2, int useSeed2) {
  s->state = 0x79dedea3u * (seed1 ^ (uint32_t)(saru_s(seed1) >> 14));
  if (useSeed2) s->wstate = (s->state + seed2) ^ (uint32_t)(saru_s(s->state) >> 8);
  else s->wstate = seed1 ^ (uint32_t)(saru_s(s->state) >> 8);
  s->state = s->state + (s->wstate * (s->wstate ^ 0xdddf97f5u));
  s->wstate = 0xABCB96F7u + (s->wstate >> 1);
}
static inline Saru saru1(uint32_t seed) { /* :222-228 */
  Saru s;
  saru_finish_seed(&s, seed, 0, 0);
  return s;
}
static inline Saru saru2(uint32_t seed1, uint32_t seed2) { /* :232-248 */
  seed2 += seed1 << 16;
  seed1 += seed2 << 11;
  seed2 += (uint32_t)(saru_s(seed1) >> 7);
  seed1 ^= (uint32_t)(saru_s(seed2) >> 3);
  seed2 *= 0xA5366B4Du;
  seed2 ^= seed2 >> 10;
  seed2 ^= (uint32_t)(saru_s(seed2) >> 19);
  seed1 += seed2 ^ 0x6d2d4e11u;
  Saru s;
  saru_finish_seed(&s, seed1, seed2, 1);
  return s;
}
static inline Saru saru3(uint32_t seed1, uint32_t seed2, uint32_t seed3) { /* :253-273 */
  seed3 ^= (seed1 << 7) ^ (seed2 >> 6);
  seed2 += (seed1 >> 4) ^ (seed3 >> 15);
  seed1 ^= (seed2 << 9) + (seed3 << 8);
  seed3 ^= 0xA5366B4Du * ((seed2 >> 11) ^ (seed1 << 1));
  seed2 += 0x72BE1579u * ((seed1 << 4) ^ (seed3 >> 16));
  seed1 ^= 0X3F38A6EDu * ((seed3 >> 5) ^ (uint32_t)(saru_s(seed2) >> 22));
  seed2 += seed1 * seed3;
  seed1 += seed3 ^ (seed2 >> 2);
  seed2 ^= (uint32_t)(saru_s(seed2) >> 17);
  Saru s;
  saru_finish_seed(&s, seed1, seed2, 1);
  return s;
}
static inline uint32_t saru_u32(Saru *s) { /* :340-352 with steps=1 */
  s->state = SARU_LCGA * s->state + SARU_LCGC;                                   /* advanceLCG<1> */
  s->wstate = s->wstate + SARU_oWeylOffset +
              ((uint32_t)(saru_s(s->wstate) >> 31) & SARU_oWeylPeriod);          /* advanceWeyl<1> :215-218 */
  uint32_t v = (s->state ^ (s->state >> 26)) + s->wstate;
  return (v ^ (v >> 20)) * 0x6957f5a7u;
}
static inline float saru_f(Saru *s) { /* :360-364 */
  return ((int32_t)(saru_u32(s) >> 1)) * (1.0f / 0x80000000);
}
static inline float saru_f_range(Saru *s, float low, float high) { /* :368-373 */
  const float TWO_N32 = 0.232830643653869628906250e-9f;
  /* a*b + c with c itself a product (0.5f*(high+low)): pinned as FMA(a, b, c) */
  return fmaf((float)((int32_t)saru_u32(s)), TWO_N32 * (high - low), 0.5f * (high + low));
}
/* gf: :115-128.  Returns (r*sin, r*cos)*std + mean. */
static inline void saru_gf(Saru *s, float mean, float std, float *o0, float *o1) {
  const float pi2 = (float)(2.0 * M_PI);
  float u0;
  do { u0 = saru_f(s); } while (u0 <= FLT_MIN);
  const float u1 = saru_f(s);
  const float r = sqrtf(-2.0f * logf(u0));
  const float theta = pi2 * u1;
  *o0 = fmaf(r * sinf(theta), std, mean);
  *o1 = fmaf(r * cosf(theta), std, mean);
}
/* gd: :130-143.  The reference's "double" Gaussian still draws float uniforms and uses the float
 * libm entry points; only the final scale is in double. */
static inline void saru_gd(Saru *s, double mean, double std, double *o0, double *o1) {
  const double pi2 = 2.0 * M_PI;
  double u0;
  do { u0 = saru_f(s); } while (u0 <= DBL_MIN);
  const double u1 = saru_f(s);
  const double r = sqrtf((float)(-2.0 * logf((float)u0)));
  const double theta = pi2 * u1;
  *o0 = r * sinf((float)theta) * std + mean;
  *o1 = r * cosf((float)theta) * std + mean;
}
#endif
