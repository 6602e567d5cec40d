// Saru counter-style PRNG on the device: three-seed hash constructor -> LCG + offset Weyl sequence ->
// 32-bit output hash.  The integer stream must equal the reference's bit for bit because every
// stochastic kernel keys its noise as Saru(id, step, seed) (third_party/saruprng.cuh:222-273 seeding,
// :340-352 output, :360-386 float conversion, :115-128 Box-Muller).
#pragma once
#include "device_common.hpp"

namespace uammd_hip {

struct Saru {
  uint state, wstate;
  static constexpr uint LCGA = 0x4beb5d59u, LCGC = 0x2600e1f7u;
  static constexpr uint oWeylPeriod = 0xda879addu, oWeylOffset = 0x8009d14bu;

  UH_HD static int s(uint x) { return (int)x; }
  UH_HD void finish(uint seed1, uint seed2, bool two) {
    state = 0x79dedea3u * (seed1 ^ (uint)(s(seed1) >> 14));
    wstate = two ? ((state + seed2) ^ (uint)(s(state) >> 8)) : (seed1 ^ (uint)(s(state) >> 8));
    state = state + (wstate * (wstate ^ 0xdddf97f5u));
    wstate = 0xABCB96F7u + (wstate >> 1);
  }
  UH_HD explicit Saru(uint seed) { finish(seed, 0u, false); }
  UH_HD Saru(uint seed1, uint seed2) {
    seed2 += seed1 << 16;
    seed1 += seed2 << 11;
    seed2 += (uint)(s(seed1) >> 7);
    seed1 ^= (uint)(s(seed2) >> 3);
    seed2 *= 0xA5366B4Du;
    seed2 ^= seed2 >> 10;
    seed2 ^= (uint)(s(seed2) >> 19);
    seed1 += seed2 ^ 0x6d2d4e11u;
    finish(seed1, seed2, true);
  }
  UH_HD Saru(uint seed1, uint seed2, uint seed3) {
    seed3 ^= (seed1 << 7) ^ (seed2 >> 6);
    seed2 += (seed1 >> 4) ^ (seed3 >> 15);
    seed1 ^= (seed2 << 9) + (seed3 << 8);
    seed3 ^= 0xA5366B4Du * ((seed2 >> 11) ^ (seed1 << 1));
    seed2 += 0x72BE1579u * ((seed1 << 4) ^ (seed3 >> 16));
    seed1 ^= 0X3F38A6EDu * ((seed3 >> 5) ^ (uint)(s(seed2) >> 22));
    seed2 += seed1 * seed3;
    seed1 += seed3 ^ (seed2 >> 2);
    seed2 ^= (uint)(s(seed2) >> 17);
    finish(seed1, seed2, true);
  }
  UH_HD uint u32() {
    state = LCGA * state + LCGC;
    wstate = wstate + oWeylOffset + ((uint)(s(wstate) >> 31) & oWeylPeriod);
    const uint v = (state ^ (state >> 26)) + wstate;
    return (v ^ (v >> 20)) * 0x6957f5a7u;
  }
  UH_HD float f() { return ((int)(u32() >> 1)) * (1.0f / 0x80000000); }
  UH_HD float f(float low, float high) {
    const float TWO_N32 = 0.232830643653869628906250e-9f;
    return fmaf((float)((int)u32()), TWO_N32 * (high - low), 0.5f * (high + low));
  }
  // Box-Muller pair N(mean, std); transcendental: not bit-reproducible against libm.
  UH_HD float2 gf(float mean, float std) {
    const float pi2 = 6.283185307179586f;
    float u0;
    do { u0 = f(); } while (u0 <= 1.17549435e-38f);
    const float u1 = f();
    const float r = sqrtf(-2.0f * logf(u0));
    const float theta = pi2 * u1;
    return make_float2(fmaf(r * sinf(theta), std, mean), fmaf(r * cosf(theta), std, mean));
  }
  // The same pair from the same two uniforms with the hardware transcendentals: log2 and sqrt to ~1 ulp (v_log_f32, v_sqrt_f32) and
  // sin / cos of 2 pi u1 from v_sin_f32 / v_cos_f32, whose argument is in revolutions (no 2 pi rounding, one instruction each; the
  // library's sincospif was a third of the FCM node operator: C4 solve 0.212 -> 0.205 ms, the T = 1 error against the oracle unchanged
  // at 3.2e-7).  Differs from gf() by a few 1e-7 of the standard deviation — the same size as gf()'s own distance from the host
  // libm; used where the draw is the hot loop (the 6 normals per Fourier node of the FCM noise, FCM_impl.cuh:466-476).
  UH_HD float2 gf_fast(float mean, float std) {
#if defined(__HIP_DEVICE_COMPILE__)
    float u0;
    do { u0 = f(); } while (u0 <= 1.17549435e-38f);
    const float u1 = f();
    const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u0));  // sqrt(-2 ln u0), ln = log2 * ln 2
    // v_sin_f32 / v_cos_f32 take their argument in revolutions: sin(2 pi u1) is one instruction (the library's sincospif is ~50)
    const float sn = __builtin_amdgcn_sinf(u1), cs = __builtin_amdgcn_cosf(u1);
    return make_float2(fmaf(r * sn, std, mean), fmaf(r * cs, std, mean));
#else
    return gf(mean, std);  // (host pass of the compiler only)
#endif
  }
};

}  // namespace uammd_hip
